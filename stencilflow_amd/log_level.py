"""Verbosity levels accepted by ``KernelChainGraph`` and ``run_program``
(same names and ordering as reference stencilflow/log_level.py:15-24)."""

import enum
import functools


@functools.total_ordering
class LogLevel(enum.Enum):
    NO_LOG = 0
    BASIC = 1
    MODERATE = 2
    FULL = 3

    def __lt__(self, other):
        if self.__class__ is other.__class__:
            return self.value < other.value
        return NotImplemented
