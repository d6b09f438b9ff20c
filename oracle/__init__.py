"""ORACLE -- test infrastructure only (see oracle/lstt_ref.py header).

CPU restatement of the reference hot path.  Never imported by ``rmem_amd``.
"""
