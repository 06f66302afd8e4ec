import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_tables():
    with open(os.path.join(GOLD, "tables.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_games():
    """60 reference games (uniform / beetle-happy / queen-attacking playouts) + 14 with voluntary passes."""
    games = []
    for name in ("games_full.json.gz", "games_pass.json.gz"):
        with gzip.open(os.path.join(GOLD, name), "rt") as f:
            games += json.load(f)["games"]
    return games
