"""`src` alias package: the reference's scripts import src.features / src.data / src.utils."""
