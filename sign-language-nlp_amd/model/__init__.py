"""Drop-in for the reference's ``model`` package (/root/reference/model/__init__.py:1-3):
``pydoc.locate("model.Transformer")`` etc. (helper.py:93) resolve to the HIP-backed classes."""
from . import util  # noqa: F401
from .encoder_decoder_attn import EncoderDecoderGRUAttn, EncoderDecoderLSTMAttn  # noqa: F401
from .transformer import Transformer  # noqa: F401
