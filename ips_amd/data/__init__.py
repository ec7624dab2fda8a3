"""Datasets on either side of the hot path (mirror of the reference's ``data`` package, Megapixel MNIST)."""
