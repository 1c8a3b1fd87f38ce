"""mickey_amd: MI355X-native implementation of the MicKey inference hot path
(DINOv2 encoder + heads -> dual-softmax matcher -> probabilistic-Procrustes RANSAC)."""
__version__ = "0.1.0"
