"""Command line front end: ``python -m photonbend_amd {make-photo, alter-photo, make-pano}``."""
