"""``pickle_module`` for ``torch.save`` / ``torch.load`` of the ``{'model', 'mask', 'args'}`` checkpoint
(nesvor/cli/io.py:33-59) that keeps the file interchangeable with the reference:

* writing: the container classes of this package are recorded under the REFERENCE's import paths
  (``nesvor.image.image.Volume``, ``nesvor.transform.transform.RigidTransform`` ...), so a checkpoint written here
  unpickles in an environment that has the reference installed.  The two packages keep the same instance attributes
  (``image / mask / transformation / resolution_*`` and ``trans_first / _axisangle / _matrix``), and pickle restores
  plain objects by ``cls.__new__`` + ``__dict__``, so nothing else is needed;
* reading: ``nesvor.image*`` / ``nesvor.transform*`` globals found in a file resolve to this package's classes, so a
  checkpoint written by the reference loads here without the reference being importable.

Tensors never pass through these classes: torch stores them as zip records via ``persistent_id``.
"""
import pickle as _pickle
from pickle import *  # noqa: F401,F403  (torch.load / torch.save look attributes up on the module)

_TO_REFERENCE = {
    ("nesvor_amd.image", "Image"): ("nesvor.image.image", "Image"),
    ("nesvor_amd.image", "Slice"): ("nesvor.image.image", "Slice"),
    ("nesvor_amd.image", "Volume"): ("nesvor.image.image", "Volume"),
    ("nesvor_amd.image", "Stack"): ("nesvor.image.image", "Stack"),
    ("nesvor_amd.transform", "RigidTransform"): ("nesvor.transform.transform", "RigidTransform"),
}
_FROM_REFERENCE = {"nesvor.image": "nesvor_amd.image", "nesvor.transform": "nesvor_amd.transform"}


class Pickler(_pickle._Pickler):
    """Pure-Python pickler (the C one cannot be hooked at class-reference level); only the small object graph of a
    checkpoint goes through it."""

    def save_global(self, obj, name=None):
        key = (getattr(obj, "__module__", None), getattr(obj, "__qualname__", None))
        if key in _TO_REFERENCE:
            module, qualname = _TO_REFERENCE[key]
            self.write(_pickle.GLOBAL + module.encode("ascii") + b"\n" + qualname.encode("ascii") + b"\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)


class Unpickler(_pickle.Unpickler):
    def find_class(self, module, name):
        for prefix, ours in _FROM_REFERENCE.items():
            if module == prefix or module.startswith(prefix + "."):
                import importlib

                target = importlib.import_module(ours)
                if hasattr(target, name):
                    return getattr(target, name)
        return super().find_class(module, name)


def dump(obj, file, protocol=None, **kw):
    Pickler(file, protocol, **kw).dump(obj)


def load(file, **kw):
    return Unpickler(file, **kw).load()
