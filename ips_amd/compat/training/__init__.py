"""`training` package name of the reference, resolved to ips_amd.training."""
