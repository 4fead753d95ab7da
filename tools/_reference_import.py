"""Import the read-only reference (/root/reference/src) in THIS container only.

Test/golden-generation infrastructure: never imported by the product package, never shipped.
The reference's optional third-party imports (torchvision, pretrainedmodels, semantic_version,
pose3d_utils, ...) are absent from this image; a meta-path finder fabricates inert stand-ins for
those *module names only* so that `margipose.models.margipose_model` can be imported in place.
No reference source is copied: the files are executed from /root/reference.
"""
import importlib.abc
import importlib.machinery
import sys
import types
from unittest import mock

REFERENCE_SRC = '/root/reference/src'
_STUB_ROOTS = {'semantic_version', 'torchvision', 'pretrainedmodels', 'pose3d_utils',
               'importlib_resources', 'h5py', 'torchdata', 'tele', 'sacred', 'pyshowoff'}


class _InertLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        m.__getattr__ = lambda name: mock.MagicMock(name=name)
        return m

    def exec_module(self, module):
        pass


class _InertFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.')[0] in _STUB_ROOTS or fullname == 'torch._six':
            return importlib.machinery.ModuleSpec(fullname, _InertLoader(), is_package=True)
        return None


def import_reference():
    """Returns (dsntnn_module, margipose_model_module, CanonicalSkeletonDesc)."""
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    if not any(isinstance(f, _InertFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _InertFinder())
    import semantic_version
    semantic_version.Version = str

    class _Spec:
        def __init__(self, s):
            self.s = s

        def __contains__(self, v):
            return True
    semantic_version.Spec = _Spec
    import torch._six as six
    six.string_classes = (str,)
    six.int_classes = (int,)
    import margipose.dsntnn as dsntnn
    import margipose.models.margipose_model as mm
    from margipose.data.skeleton import CanonicalSkeletonDesc
    return dsntnn, mm, CanonicalSkeletonDesc
