from ips_amd.architecture.ips_net import IPSNet  # noqa: F401
