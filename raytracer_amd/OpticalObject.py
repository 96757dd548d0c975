"""An optical element = a behaviour (`Type`) attached to a shape (`Geometry`)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any


@dataclass
class OpticalObject:
    Type: Any
    Geometry: Any
