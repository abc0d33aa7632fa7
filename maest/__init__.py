"""Alias of maest_amd under the reference's public package name (reference: models/__init__.py:1,
pyproject.toml:35-38 maps ``maest`` -> ``models/``), so ``from maest import get_maest`` keeps working."""
from maest_amd import MAEST, get_maest  # noqa: F401
