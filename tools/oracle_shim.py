"""Import the reference's ``model`` package in the BUILD CONTAINER ONLY.

Build-owned helper.  ``/root/reference`` is read-only
and does not exist on the GPU box; nothing under tests/, bench.py or the
product imports this file.  A bare ``import model`` fails on the reference's
``dataset`` package (needs skorch/torchtext, absent), so a stub ``dataset``
package exposing only the real ``dataset.constant`` literals is registered
first.
"""
import importlib.util
import sys
import types

REF = "/root/reference"


def import_reference_model():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "dataset" not in sys.modules:
        pkg = types.ModuleType("dataset")
        pkg.__path__ = []
        sys.modules["dataset"] = pkg
        spec = importlib.util.spec_from_file_location(
            "dataset.constant", f"{REF}/dataset/constant/__init__.py",
            submodule_search_locations=[f"{REF}/dataset/constant"])
        const = importlib.util.module_from_spec(spec)
        sys.modules["dataset.constant"] = const
        spec.loader.exec_module(const)
        pkg.constant = const
    import model  # noqa: the reference package
    assert model.__file__.startswith(REF), model.__file__
    return model


class Vocab:
    """torchtext-0.6 stand-in: ``<unk>``=0, ``<pad>``=1; unknown keys
    (e.g. ``<bos>`` on the target field) resolve to 0 like its defaultdict."""

    def __init__(self, n):
        self.itos = ["<unk>", "<pad>"] + [f"t{i}" for i in range(n - 2)]

        class _S(dict):
            def __missing__(s, k):
                return 0

        self.stoi = _S({t: i for i, t in enumerate(self.itos)})

    def __len__(self):
        return len(self.itos)
