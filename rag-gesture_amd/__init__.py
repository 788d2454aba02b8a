"""MI355X-native RAG-Gesture inference hot path."""
from . import synth  # noqa: F401
