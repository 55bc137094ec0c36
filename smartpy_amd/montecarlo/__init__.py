"""Monte-Carlo workflows on the GPU engine (counterpart of smartpy/montecarlo): LHS, GLUE, Best, Total."""
from .lhs import LHS
from .glue import GLUE
from .best import Best
from .total import Total

__all__ = ['LHS', 'GLUE', 'Best', 'Total']
