"""Drop-in for the reference's `networks` package on the hot path (networks/__init__.py:1-9)."""
from .resnet_encoder import ResnetEncoder
from .depth_decoder import DepthDecoder
from .pose_decoder import PoseDecoder
from .pose_cnn import PoseCNN
from .fusion import Fusion_v3, FeatureFusionBlock_v3, ResidualAttentionUnit, AttentionConv, UpscalePS
from .convgru import ConvGRUBlocks_v5, ConvGRUModel_v1, ConvGRUCell
