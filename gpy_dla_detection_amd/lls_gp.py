"""
GP model with intervening Lyman-limit systems: Lyman-series Voigt lines plus the Lyman-limit break.

The reference ships this profile in `gpy_dla_detection/voigt_lls.py` (:258-360) and the model class
in `examples/gp_find_lls.py:124-224` (`LLSGPDR12`, a `DLAGP` whose `this_dla_gp` calls the LLS
profile).  Here it is the same sweep with one flag on the device context
(`DLAGP_FLAG_LLS_BREAK`): every absorber's exponent gains
`N / 10^17.2 (lambda_rest / 911.7641 Å)^3` bluewards of the limit.  The sweep runs on the MFMA kernel
(one absorber per sample at level 1, cached profiles at the deeper levels).
"""
from .dla_gp import DLAGP


class LLSGP(DLAGP):
    """`DLAGP` with the Lyman-limit continuum opacity in the absorption profile; same constructor."""

    lls_break = True
