# Namespace bridge (see source/__init__.py): modules this package does not provide (source.base.utils, mesh_io,
# file_utils, point_cloud, ...) come from the reference checkout's source/base on sys.path.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
