from .Rotations import Rotations
