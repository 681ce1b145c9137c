"""hommx_amd -- MI355X-native batched micro-cell solver behind the HOMMX solver-class API."""

from .batch import MicroCellPlan  # noqa: F401

__all__ = ["MicroCellPlan"]
