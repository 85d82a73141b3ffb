"""Importable alias of the product package.

The sources live in `variance-aware-weight_amd/` (the directory name the project layout
prescribes); a hyphen cannot appear in a Python import, so this stub only points the
package path at that directory and re-exports its public names.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "variance-aware-weight_amd")
__path__.append(_real)

from ._api import *  # noqa: F401,F403,E402
