"""swin_v2_weather_amd: MI355X-native hot path of NERSC/swin_v2_weather."""
__version__ = "0.1.0"
