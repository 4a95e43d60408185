"""longsom_amd — MI355X-native implementation of LongSom's SComatic-derived SNV hot path."""
__version__ = "0.1.0"
