"""Bootstrap shared by the drop-in shims: puts the repo root (the directory that holds `pistoseg_amd/`) on sys.path, so that
`PYTHONPATH=<repo>/pistoseg_amd/compat` alone is enough for the reference's stage scripts to resolve their
`from models.* import ...` / `from loss import ...` / `import utils` lines to the MI355X mirrors."""
import importlib.util
import os
import sys

COMPAT_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(os.path.dirname(COMPAT_DIR))
if REPO_ROOT not in sys.path:
    sys.path.append(REPO_ROOT)


def overlay_next_on_path(module_name: str, namespace: dict) -> bool:
    """The reference tree has sibling top-level modules of the same name (`utils.py`: label parsing, background detection,
    visualisation; `loss.py`: an unused DiceLoss) whose host-side helpers the stage scripts and `dataset.py` keep using.  Execute
    the NEXT `<module_name>.py` found on sys.path after this directory into `namespace`, so that the shim only replaces the names
    it re-defines afterwards.  Returns False when there is none (stand-alone use: the shim's own names are all there is)."""
    for d in sys.path:
        if not d or os.path.abspath(d) == COMPAT_DIR:
            continue
        cand = os.path.join(d, module_name + ".py")
        if os.path.isfile(cand):
            spec = importlib.util.spec_from_file_location(f"_reference_{module_name}", cand)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            namespace.update({k: v for k, v in vars(mod).items() if not k.startswith("__")})
            namespace["__reference_file__"] = cand
            return True
    return False
