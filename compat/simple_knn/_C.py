from d3ga_amd.tetra import distCUDA2  # noqa: F401
