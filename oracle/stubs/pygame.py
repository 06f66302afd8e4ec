"""Test-harness stand-in for pygame (absent in this image, no network).

The reference's headless path only *constructs* ``pg.Rect`` objects
(inventory_frame.py:29-32,48; turn_panel.py:38-39); nothing on the arithmetic
path reads them.  This file is NOT part of the product: it is only put on
PYTHONPATH by oracle/gen_golden.py when importing /root/reference.
"""


class Rect:
    def __init__(self, *args, **kwargs):
        self.args = args
