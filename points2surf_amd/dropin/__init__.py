"""Drop-in import root.  Put this directory in front of the reference checkout on ``sys.path``
(``PYTHONPATH=<repo>/points2surf_amd/dropin:<reference>``): ``import source.points_to_surf_eval`` and
``import source.points_to_surf_model`` then resolve to the MI355X engine, every other ``source.*``
module (sdf, base.evaluation, ...) still resolves to the reference.  See INTEGRATION.md."""
