"""Condition encoders of the reference's lidargen/models/unets/encoders/ that are on the path:
`object_gen_encoder.ObjectGenEncoder` (+ its Fourier `embedder`) of the foreground-object branch."""
