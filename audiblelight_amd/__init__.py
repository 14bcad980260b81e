"""MI355X-native synthesis hot path with AudibleLight's function boundary."""
__version__ = "0.1.0"
