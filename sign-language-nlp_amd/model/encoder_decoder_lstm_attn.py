"""Import path kept for ``pydoc.locate("model.EncoderDecoderLSTMAttn")`` / ``from model.encoder_decoder_lstm_attn import ...``;
the class itself is generated next to its base (encoder_decoder_attn.py)."""
from .encoder_decoder_attn import EncoderDecoderLSTMAttn  # noqa: F401
