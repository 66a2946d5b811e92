"""qcqp_amd -- MI355X-native Suggest-and-Improve engine behind the cvxgrp/qcqp API.

``from qcqp_amd import *`` yields the same names as ``from qcqp import *`` in the reference
(qcqp/__init__.py:27-29): QCQP, RANDOM, SPECTRAL, SDR, COORD_DESCENT, ADMM, DCCP, IPOPT.
"""
from .settings import RANDOM, SPECTRAL, SDR, COORD_DESCENT, ADMM, DCCP, IPOPT  # noqa: F401
from .form import QuadraticFunction, QCQPForm  # noqa: F401


def __getattr__(name):
    # the API facade and the engine need the HIP library; import them lazily so that the pure
    # host modules (problems, form, settings) stay importable on a machine without it.
    if name in ('QCQP', 'Variable', 'Problem'):
        from . import api
        return getattr(api, name)
    if name in ('Engine', 'EngineError'):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)


__all__ = ['QCQP', 'RANDOM', 'SPECTRAL', 'SDR', 'COORD_DESCENT', 'ADMM', 'DCCP', 'IPOPT']
