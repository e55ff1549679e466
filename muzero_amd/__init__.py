"""muzero_amd — MI355X-native MuZero self-play search (hot path of DHDev0/Muzero)."""
__version__ = "0.1.0"
