from .backbone.swin import D2SwinTransformer
from .backbone.dinat import D2DiNAT
from .pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
from .pixel_decoder.fpn import build_pixel_decoder
from .pixel_decoder.transdssl import TransDSSL
from .transformer_decoder.oneformer_transformer_decoder import ContrastiveMultiScaleMaskedTransformerDecoder
from .meta_arch.oneformer_head import OneFormerHead
