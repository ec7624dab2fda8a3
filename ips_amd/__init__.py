"""ips_amd - MI355X (gfx950) implementation of the Iterative Patch Selection hot path.

``ips_amd.architecture`` mirrors the reference's ``architecture`` package
(``IPSNet``, ``Transformer``, ...); ``ips_amd.hip`` binds the C ABI of
``include/ipsx.h`` (``ips_amd/lib/libipsx.so``, built from ``ips_amd/csrc``).
"""

__version__ = "0.1.0"
