"""pixel-embedded-affinity_amd -- the MI355X-native embedding -> affinity hot path.

A drop-in for the one data-parallel hot path of weih527/Pixel-Embedded-Affinity: per-pixel embedding
-> K-offset affinity map -> class-balanced weighted-MSE loss -> gradient, as hand-written HIP for
gfx950 behind a C ABI (include/pea.h), with this thin Python layer mirroring the reference's own
function names so its main.py / inference.py call sites only change their import line
(see INTEGRATION.md).

Import name: the directory is not a valid Python identifier, so load it through
`__graft_entry__.load_package()` (registers it as `pixel_embedded_affinity_amd`).
"""
from . import _lib
from ._lib import PeaLibraryError, build
from .affinity_op import (AffinityMap, AffinitySpec, FusedAffinityMSE, Graphed, LabelsAffinityMSE, affinity_infer, backward, check_label_ranges,
                          graphed)
from .loss.loss import WeightedMSE
from .loss.loss_embedding_mse import (ema_embedding_loss, ema_embedding_loss_from_labels, embedding2affs, embedding_loss,
                                      embedding_loss_from_labels)
from .loss.loss_embedding_mse_3d import (ema_embedding_loss_norm1, ema_embedding_loss_norm5, ema_embedding_loss_norm5_from_labels,
                                         ema_embedding_loss_norm6, embedding_loss_norm6,
                                         embedding_loss_norm1, embedding_loss_norm1_from_labels, embedding_loss_norm5,
                                         embedding_loss_norm5_from_labels, inf_embedding_loss_norm1, inf_embedding_loss_norm5)
from .utils.affinity_ours import gen_offsets, multi_offset
from .utils.postproc import fill_border_relu_, relu_
from .utils.targets import gen_affs_ours, gen_targets, seg_to_aff
from .harness.stitch import VolumeStitcher
from .harness.handoff import AffsCollector
from .harness.head_loss import HeadAffinityMSE, head_embedding_loss
from .harness.train_step import CvpppTrainStep, convert_consistency_flip, label_pyramid, make_optimizer
from .model.unet2d_residual import ResidualUNet2D_deep
from .model.head import EmbeddingHead, OutConv, head_conv3d_block
from .harness.loss_section import (ac3ac4_loss_section, ac3ac4_loss_section_composed, ac3ac4_loss_section_from_labels,
                                   cvppp_loss_section, cvppp_loss_section_composed,
                                   cvppp_loss_section_from_labels, cvppp_label_weight_tables, cvppp_validation_section,
                                   deep_weight_factor,
                                   finish_pred_2d_, finish_pred_3d_)

__all__ = [
    "PeaLibraryError", "build", "backward", "graphed", "Graphed", "check_label_ranges", "AffinityMap", "AffinitySpec", "FusedAffinityMSE", "affinity_infer", "WeightedMSE",
    "embedding_loss", "ema_embedding_loss", "embedding2affs", "embedding_loss_norm1", "embedding_loss_norm5",
    "ema_embedding_loss_norm1", "ema_embedding_loss_norm5", "inf_embedding_loss_norm1", "inf_embedding_loss_norm5",
    "gen_offsets", "multi_offset", "fill_border_relu_", "relu_", "cvppp_loss_section", "ac3ac4_loss_section",
    "deep_weight_factor", "finish_pred_2d_", "finish_pred_3d_", "gen_targets", "gen_affs_ours", "seg_to_aff", "VolumeStitcher", "embedding_loss_from_labels",
    "ema_embedding_loss_from_labels", "LabelsAffinityMSE", "cvppp_loss_section_from_labels", "cvppp_loss_section_composed", "ac3ac4_loss_section_composed",
    "ac3ac4_loss_section_from_labels",
    "embedding_loss_norm1_from_labels", "embedding_loss_norm5_from_labels", "ema_embedding_loss_norm5_from_labels",
    "embedding_loss_norm6", "ema_embedding_loss_norm6", "EmbeddingHead", "OutConv", "head_conv3d_block", "cvppp_label_weight_tables", "cvppp_validation_section",
]
