"""Inert stand-in for the `tensorflow` module, used ONLY by oracle/gen_golden.py in the dev
container so that the reference's non-TensorFlow code (Dataset, Sampler, Evaluation) can be
imported to generate golden vectors. It implements nothing: every attribute
access or call returns another inert object.  It is test infrastructure, never shipped or imported
by the product."""


class _Any:
    def __getattr__(self, name):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


config = random = keras = losses = nn = initializers = math = _Any()


def is_tensor(x):
    return False
