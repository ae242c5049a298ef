"""Drop-in `simple_knn` package surface (models/mesh_net.py:22 does `from simple_knn._C import distCUDA2`)."""
