"""Constants of src/func_util/math_func.py:22-31."""
LOG_NUM_STAB = 2 ** (-16)
PROBA_MIN = LOG_NUM_STAB
PROBA_MAX = 1.0
LOG_VAR_MAX = 10.
LOG_VAR_MIN = -18.4207
