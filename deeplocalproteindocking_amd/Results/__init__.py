from .DockerParser import DockerParser, kabsch_rmsd
from . import InterfaceSelection
