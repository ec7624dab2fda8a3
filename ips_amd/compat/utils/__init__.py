"""`utils` package name of the reference, resolved to ips_amd.utils (see ../architecture/__init__.py)."""
