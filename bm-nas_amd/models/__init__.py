"""`models` — host-side mirror of the reference's package of the same name.

Only the fusion-search hot path and its callers live here.  Everything the reference keeps
elsewhere under `models/` (unimodal backbones in `models/central`, `models/utils.py`,
`models/auxiliary/inflated_resnet.py`, ...) is out of scope and is picked up from the reference
checkout when that is ALSO on sys.path (after this directory): the package path is extended,
this directory wins for every module it provides.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
