"""Skeleton ``stemseg`` package: only reached when no sabarim/STEm-Seg checkout is importable (see ../README.md)."""
