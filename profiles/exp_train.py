import os, sys, time, importlib, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
pkg = ge.load_package()
mod = importlib.import_module(ge.PKG_NAME + ".model.unet2d_residual")
ts = importlib.import_module(ge.PKG_NAME + ".harness.train_step")
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
def run(bench, b=2, steps=8):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(555)
    net = mod.ResidualUNet2D_deep(in_channels=3, out_channels=2, nfeatures=[16, 32, 64, 128, 256], emd=16).to(dev)
    stepper = ts.CvpppTrainStep(net, ts.make_optimizer(net))
    g = torch.Generator(device=dev).manual_seed(1000)
    x = torch.randn(b, 3, 544, 544, generator=g, device=dev); x_ema = x + 0.1 * torch.randn(b, 3, 544, 544, generator=g, device=dev)
    labels = torch.from_numpy(synth.synth_labels(b, (1, 544, 544), 555)[:, 0].copy()).to(dev).to(torch.int32)
    for _ in range(4): stepper.step(x, x_ema, labels)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): loss = stepper.step(x, x_ema, labels)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return b * steps / dt, dt / steps * 1e3, float(loss)
import os
for bench in ((False,) if os.environ.get("ONE") else (False, True)):
    for b in ((2,) if os.environ.get("ONE") else (2, 8)):
        print("benchmark=%s b=%d: %.1f img/s, %.2f ms/step, loss %.4f" % ((bench, b) + run(bench, b)), flush=True)
