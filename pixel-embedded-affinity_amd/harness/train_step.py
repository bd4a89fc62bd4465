"""One CVPPP training step, the way one rank of a multi-GPU job runs it (BASELINE.json "train imgs/sec").

What the reference does per iteration (scripts_cvppp/main.py:230-319, shipped cvppp.yaml):
    model(inputs) -> five embeddings + mask logits; model(ema_inputs) -> the EMA embedding (sharing_weights: the same net,
    train mode) -> convert_consistency_flip (per-sample un-flip, DETACHED: scripts_cvppp/data/data_consistency.py:34-45) ->
    five self losses + the EMA cross loss (:284-293) -> ct_weight (0.0) * MSE(embedding, ema_embedding) (:296) ->
    loss.backward() -> relu(pred) -> Adam(lr 1e-4, betas (0.9, 0.999), eps 0.01, weight_decay 1e-6, amsgrad) .step() (:492-493)
under nn.DataParallel with the loss on cuda:0 over the gathered batch (:117-125, :266-293).

Here: one process per GPU.  The backbone (model/unet2d_residual.py, plain PyTorch-ROCm, its heads on pea_head_*) is wrapped by
the caller in DistributedDataParallel (backend "nccl" = RCCL over xGMI): its bucketed gradient all-reduce overlaps the
backward and averages over ranks, which with the LOCAL normaliser 1/(b*W) of each rank's loss reproduces the reference's
1/(B*W) exactly (utils/shard.py).  The loss section is this package's labels-in section (harness/loss_section.py): targets,
masks and class-balance weights are evaluated from the int32 label pyramid inside the kernels, so the per-step host->device
traffic is the images and five label maps instead of ~40 MB of target / weight / mask tensors per image.
The second forward runs under no_grad: its output is detached by convert_consistency_flip, so the graph the reference
builds for it is never used (BatchNorm statistics update the same way)."""
import numpy as np
import torch
import torch.nn.functional as F

from ..loss.loss import WeightedMSE
from ..utils.affinity_ours import multi_offset
from .loss_section import cvppp_loss_section, cvppp_loss_section_from_labels


def convert_consistency_flip(ema_embedding, rules):
    """per-sample inverse of the EMA branch's flips (rules[b] = (x-flip, y-flip, xy-transpose), drawn by the data provider);
    returns a DETACHED tensor, as the reference's convert_consistency_flip does (data_consistency.py:36)"""
    out = ema_embedding.detach().clone()
    if rules is None:
        return out
    # one host copy of the whole table, cast the way the reference does (rules.data.cpu().numpy().astype(np.uint8), :37)
    r = (rules.detach().cpu().numpy() if torch.is_tensor(rules) else np.asarray(rules)).astype(np.uint8)
    parts = []
    for b in range(out.shape[0]):
        t = out[b]
        if r[b][2]:
            t = t.transpose(-1, -2)
        if r[b][1]:
            t = t.flip(-2)
        if r[b][0]:
            t = t.flip(-1)
        parts.append(t)
    return torch.stack(parts, dim=0)  # (a transposed view cannot be written back over its own storage)


def label_pyramid(labels):
    """the four nearest-neighbour downsampled label maps the data provider builds on the host with
    cv2.resize(label, (0, 0), fx=fy=1/2 .. 1/16, interpolation=cv2.INTER_NEAREST) (scripts_cvppp/data/data_provider.py:199-208), on the
    device: cv2's nearest rule is src = floor(dst / fx) = 2^j * dst with an output extent of cvRound(n * fx) (round half to even),
    i.e. a strided slice trimmed to that extent (for the provider's 544 x 544 crops the slice itself)"""
    out = []
    for j in range(1, 5):
        f = 2 ** j
        ny, nx = round(labels.shape[-2] / f), round(labels.shape[-1] / f)  # Python's round() is round-half-to-even, like cvRound
        out.append(labels[..., ::f, ::f][..., :ny, :nx].contiguous())
    return out


def make_optimizer(model, base_lr=1e-4):
    return torch.optim.Adam(model.parameters(), lr=base_lr, betas=(0.9, 0.999), eps=0.01, weight_decay=1e-6, amsgrad=True)


class CvpppTrainStep(object):
    """step(inputs, ema_inputs, labels[, rules]) -> loss: forward x2, loss section, backward (DDP all-reduce inside), Adam step.
    `model` is the (DDP-wrapped) ResidualUNet2D_deep; labels int32 [b,H,W] on the model's device."""

    def __init__(self, model, optimizer=None, shifts=(1, 3, 5, 9, 27), neighbor=4, deep_weight=1, self_emb=1.0, cross_emb=1.0,
                 ct_weight=0.0, affs0_weight=1):
        self.model = model
        self.optimizer = optimizer if optimizer is not None else make_optimizer(model)
        self.offsets = multi_offset(list(shifts), neighbor)
        self.nb_half = neighbor // 2
        self.criterion = WeightedMSE()
        self.cfg = dict(deep_weight=deep_weight, self_emb=self_emb, cross_emb=cross_emb, affs0_weight=affs0_weight)
        self.ct_weight = ct_weight
        self.pred = None

    def step(self, inputs, ema_inputs, labels, rules=None):
        self.model.train()
        self.optimizer.zero_grad(set_to_none=True)
        emd4, emd3, emd2, emd1, embedding, _mask = self.model(inputs)
        with torch.no_grad():
            ema_embedding = self.model(ema_inputs)[4]
        ema_embedding = convert_consistency_flip(ema_embedding, rules)
        loss, pred, _parts = cvppp_loss_section_from_labels(embedding, [emd1, emd2, emd3, emd4], ema_embedding, labels,
                                                            label_pyramid(labels), self.criterion, self.offsets, self.nb_half,
                                                            relu_pred=True, **self.cfg)
        loss = loss + self.ct_weight * F.mse_loss(embedding, ema_embedding)  # main.py:296 (weight 0.0 in the shipped yaml)
        loss.backward()
        self.optimizer.step()
        self.pred = pred
        return loss.detach()
