"""Drop-in for the `simple_knn` package SplatLoc imports (gaussian_model.py:18)."""
