"""`from simple_knn._C import distCUDA2` (gaussian_splatting/scene/gaussian_model.py:18)."""
from splatloc_amd.knn import distCUDA2  # noqa: F401
