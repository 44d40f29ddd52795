"""Drop-in for the `simple_knn` package (reference scene/gaussian_model.py:20)."""
