# Namespace bridge: modules this package does not provide (source.sdf, source.base.*, source.data_loader,
# ...) are looked up in the other ``source`` directories on sys.path, i.e. in the reference checkout.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
