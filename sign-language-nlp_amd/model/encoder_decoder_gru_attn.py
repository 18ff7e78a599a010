"""Import path kept for ``pydoc.locate("model.EncoderDecoderGRUAttn")`` / ``from model.encoder_decoder_gru_attn import ...``;
the class itself is generated next to its base (encoder_decoder_attn.py)."""
from .encoder_decoder_attn import EncoderDecoderGRUAttn  # noqa: F401
