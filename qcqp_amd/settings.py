"""Method constants -- same names and string values as the reference (qcqp/settings.py:25-36)."""
RANDOM = "random"
SDR = "sdr"
SPECTRAL = "spectral"

suggest_methods = [RANDOM, SDR, SPECTRAL]

COORD_DESCENT = "coord-descent"
ADMM = "admm"
DCCP = "dccp"
IPOPT = "ipopt"

improve_methods = [COORD_DESCENT, ADMM, DCCP, IPOPT]
