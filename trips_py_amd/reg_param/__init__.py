"""Regularisation-parameter selection on the PROJECTED (k-sized) problem — host side, float64 NumPy/SciPy.

The reference's selectors (trips/utilities/reg_param/{gcv,discrepancy_principle,l_curve}.py) take m-length operands
(Q_A is m x k, b is m x 1).  Here every m-length contraction has already been done on the GPU (fp64-accumulated), and
the selectors see only k-sized data:
    R_A, R_L      triangular (or any) factors with R^T R = the (weighted) Gram matrices of A V and L V
    rhs           Q_A^T b  =  R_A^{-T} (A V)^T b
    resid2        ||b - Q_A Q_A^T b||^2 = ||b||^2 - ||rhs||^2      (only where the reference uses it)
"""
from .gcv import generalized_crossvalidation, gcv_function  # noqa: F401
from .discrepancy_principle import discrepancy_principle  # noqa: F401
from .l_curve import l_curve, l_curve_curvature  # noqa: F401
