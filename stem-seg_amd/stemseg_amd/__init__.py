"""stemseg_amd -- host-side mirror of STEm-Seg's embed+cluster hot path for MI355X.

Python here is host glue only (shapes, workspaces, stream plumbing, the reference-compatible class
and function names of SURVEY.md section 8(b)); every numeric stage of the decoder / clustering path
runs in the hand-written gfx950 kernels of libstemseg_hip.so (include/stemseg_hip.h).
"""
__version__ = "0.1.0"
