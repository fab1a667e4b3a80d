from kantts import _overlay

_overlay(__name__, __path__)
