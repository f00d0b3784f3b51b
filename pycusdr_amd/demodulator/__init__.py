"""Radio back ends, selected by name like the reference's ``demodulator.UHF`` / ``demodulator.STX``
modules (reference demodulator/__init__.py:3-5, demodulator_process.py:28-34)."""
from .demodulator_base import Demodulator as Demodulator_base, Operations, log  # noqa: F401
from . import UHF, STX  # noqa: F401
