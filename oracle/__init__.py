"""CPU oracle of the CPG hot path -- test infrastructure, never imported by cpg_amd/ (see ops.py)."""
from . import ops  # noqa: F401
