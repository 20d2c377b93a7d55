from dp_timeautodiff_amd.accountants import (GaussianAccountant, IAccountant, PRVAccountant, RDPAccountant,
                                             create_accountant)

from . import utils  # noqa: F401

__all__ = ["IAccountant", "GaussianAccountant", "RDPAccountant", "PRVAccountant", "create_accountant"]
