"""Gaussian-process surrogate closure (GaPFlow/models/gp.py) -- entry points used by Problem."""


def attach_surrogates(*a, **k):
    raise NotImplementedError("the GP surrogate closure is being brought up on the HIP path; "
                              "fixed-form EOS problems run today")


def make_database(input_dict):
    raise NotImplementedError("db/gp sections: GP surrogate closure not wired into Problem yet")
