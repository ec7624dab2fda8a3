"""`architecture` package name of the reference, resolved to ips_amd.

Put `<repo>/ips_amd/compat` (and `<repo>`) in front of the reference on PYTHONPATH and the
reference's unchanged `main.py` / `training/iterative.py` import the MI355X implementation:

    PYTHONPATH=<repo>/ips_amd/compat:<repo> python main.py
"""
