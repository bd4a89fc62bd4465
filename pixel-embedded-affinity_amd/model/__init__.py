from .head import EmbeddingHead, OutConv, head_conv3d_block  # noqa: F401
