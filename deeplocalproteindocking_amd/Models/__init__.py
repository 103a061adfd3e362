from .DockingModels import GlobalDockingModel, SimpleFilter
from .ProteinRepresentationModels import E3MultiResRepr4x4, SE3MultiResReprScalar, SyntheticRepr
