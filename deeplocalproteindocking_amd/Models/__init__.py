from .DockingModels import GlobalDockingModel, SimpleFilter, fused_filter_parameters, mlp_parameters
from .ProteinRepresentationModels import E3MultiResRepr4x4, SE3MultiResReprScalar, SyntheticRepr
