"""``select_model(args)`` as the reference's test driver imports it (local_test.py:14, contract of
/root/reference/src/local_train.py:19-43): the names in ``args.group`` / ``args.model`` /
``args.filter`` pick a representation plugin and a filter, both moved to the GPU.  Training itself is
out of scope of this build."""
from Models import E3MultiResRepr4x4, SE3MultiResReprScalar, SimpleFilter, SyntheticRepr

# equivariance group -> {model name -> constructor}
REPRESENTATIONS = {
    "E3": {"E3MultiResRepr4x4": lambda: E3MultiResRepr4x4(multiplier=8)},
    "SE3": {"SE3MultiResReprScalar": lambda: SE3MultiResReprScalar(multiplier=8),
            "SyntheticRepr": lambda: SyntheticRepr(num_outputs=(48,))},
}
FILTERS = {"SimpleFilter": SimpleFilter}


def select_model(args):
    models = REPRESENTATIONS.get(args.group)
    if models is None:
        raise Exception("Unknown equivariance group", args.group)
    if args.model not in models:
        raise Exception("Unknown model name", args.model)
    if args.filter not in FILTERS:
        raise Exception("Unknown filter name", args.filter)
    protein_model = models[args.model]().cuda()
    conformations_filter = FILTERS[args.filter](protein_model.get_num_outputs()).cuda()
    return protein_model, conformations_filter
