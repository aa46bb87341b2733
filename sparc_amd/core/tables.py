"""Generator tables of the reference as constants.

``MODE_CURRENT``   — modules/currents.json: machine current [A] of mode "I<n>".
``CRATER``         — modules/area_corrected.json: crater statistics, present for the odd
                     modes I1..I17 only; the reference raises ``ValueError`` for the
                     others at the first fresh spark (material.py:108-113).
"""
from __future__ import annotations

MAX_MODE = 19

MODE_CURRENT = {1: 30, 2: 35, 3: 40, 4: 50, 5: 60, 6: 68, 7: 80, 8: 95, 9: 110, 10: 130, 11: 155, 12: 180,
                13: 215, 14: 255, 15: 305, 16: 360, 17: 425, 18: 500, 19: 600}

# mode -> (ellipsoid_volume_half [um^3], ellipsoid_volume_std [um^3], depth [um])
CRATER = {
    1: (2163.6174, 447.5072, 2.9967),
    3: (2376.5835, 523.6196, 3.0443),
    5: (4866.9691, 899.0243, 3.5302),
    7: (5556.8153, 1167.296, 3.6478),
    9: (6219.2736, 1284.4152, 3.7556),
    11: (6029.8571, 1005.804, 3.7252),
    13: (8913.8402, 2949.127, 4.1537),
    15: (26468.9924, 6472.3303, 5.9966),
    17: (59549.9184, 8997.5034, 6.2647),
}

VALID_CRATER_MODES = tuple(sorted(CRATER))


def parse_mode(mode) -> int:
    """'I7' / 7 / None -> 7 / 7 / 0 (0 encodes the reference's ``None``)."""
    if mode is None:
        return 0
    if isinstance(mode, str):
        if not mode.startswith("I"):
            raise ValueError(f"current mode must look like 'I5', got {mode!r}")
        return int(mode[1:])
    return int(mode)
