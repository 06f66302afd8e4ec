"""Test-harness stand-in for ujson (absent in this image): stdlib json."""
from json import dump, dumps, load, loads  # noqa: F401
