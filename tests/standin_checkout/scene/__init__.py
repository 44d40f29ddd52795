"""Stand-in for the reference's `scene` package: only the import edge the redirect acts on (scene/__init__.py:17)."""
from scene.gaussian_model import GaussianModel  # noqa: F401
