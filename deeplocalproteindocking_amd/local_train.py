"""``select_model`` with the contract of /root/reference/src/local_train.py:19-43 (imported by
the reference's local_test.py:14): args.group / args.model / args.filter names -> (representation,
filter) on the GPU.  Training itself is out of scope."""
from Models import E3MultiResRepr4x4, SE3MultiResReprScalar, SimpleFilter, SyntheticRepr


def select_model(args):
    if args.group == 'E3':
        if args.model == "E3MultiResRepr4x4":
            protein_model = E3MultiResRepr4x4(multiplier=8).cuda()
        else:
            raise Exception("Unknown model name", args.model)
    elif args.group == 'SE3':
        if args.model == "SE3MultiResReprScalar":
            protein_model = SE3MultiResReprScalar(multiplier=8).cuda()
        elif args.model == "SyntheticRepr":
            protein_model = SyntheticRepr(num_outputs=(48,)).cuda()
        else:
            raise Exception("Unknown model name", args.model)
    else:
        raise Exception("Unknown equivariance group", args.group)

    if args.filter == "SimpleFilter":
        conformations_filter = SimpleFilter(protein_model.get_num_outputs()).cuda()
    else:
        raise Exception("Unknown filter name", args.filter)
    return protein_model, conformations_filter
