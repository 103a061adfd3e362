from .Docker import Docker
