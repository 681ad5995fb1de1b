"""dig_amd: MI355X-native (gfx950) engine for the DiG SimMIM+MoCo-v3 pre-training step."""
__version__ = "0.1.0"
