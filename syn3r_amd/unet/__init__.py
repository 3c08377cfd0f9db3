"""SVD spatio-temporal UNet on the HIP path (channels-last fp16, hand-written CDNA4 kernels)."""
