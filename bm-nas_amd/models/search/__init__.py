"""Overlay package: modules present here shadow the reference's, the rest resolve to the
reference checkout if it is on sys.path (see models/__init__.py)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
