"""Verbosity accepted by ``KernelChainGraph`` and ``run_program``.

Same member names, values and ordering as the reference's ``LogLevel``
(stencilflow/log_level.py:15-24), so ``LogLevel(2)``, ``level >= LogLevel.BASIC``
and ``.value`` behave identically; implemented as an ``IntEnum``.
"""

import enum


class LogLevel(enum.IntEnum):
    NO_LOG = 0
    BASIC = 1
    MODERATE = 2
    FULL = 3
