from .accountant import IAccountant
from .gdp import GaussianAccountant
from .prv import PRVAccountant
from .rdp import RDPAccountant

__all__ = ["IAccountant", "GaussianAccountant", "RDPAccountant", "PRVAccountant", "create_accountant"]


def create_accountant(mechanism: str) -> IAccountant:
    if mechanism == "rdp":
        return RDPAccountant()
    elif mechanism == "gdp":
        return GaussianAccountant()
    elif mechanism == "prv":
        return PRVAccountant()
    raise ValueError(f"Unexpected accounting mechanism: {mechanism}")
