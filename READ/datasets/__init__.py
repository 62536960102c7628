from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)   # the rest of this package keeps resolving to the reference checkout
