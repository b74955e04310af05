"""MI355X-native training path for 17LiQi/MultimodalSignal's CnnGruAttentionModel."""
__version__ = "0.1.0"
