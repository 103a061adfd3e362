from .DockerParser import DockerParser, kabsch_rmsd
