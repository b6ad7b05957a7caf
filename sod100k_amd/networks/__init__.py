"""Drop-in for the reference's ``CSF+Res2Net/networks`` package (csf_res2net.py, gOctConv.py)."""
