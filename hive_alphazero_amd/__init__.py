"""Importable alias of the ``hive-alphazero_amd/`` source directory.

The product package lives in ``hive-alphazero_amd/`` (a hyphen is not importable), so this
stub only extends ``__path__`` to it; every submodule (``_lib``, ``batch``, ``env_hive`` ...)
is loaded from there.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "hive-alphazero_amd"))

from ._lib import build, load, HiveError  # noqa: E402,F401
