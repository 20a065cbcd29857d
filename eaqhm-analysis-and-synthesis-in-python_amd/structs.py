"""Result types of the analysis, attribute-compatible with the reference's structs.py:7-34.

`Deterministic.__init__` stores its `amplitudes` argument under the attribute name `ak` (structs.py:14);
the driver later attaches a separate `amplitudes` attribute when it packs the results
(functions.py:409).  Both quirks are kept because callers of the reference may rely on them.
"""


class _Record:
    def __str__(self):
        return str(self.__dict__)

    __repr__ = __str__


class Deterministic(_Record):
    """One analysis instant: `ti` (0-based sample index), speech/voicing flags and — for voiced
    instants — DC term `a0`, harmonic `amplitudes`, instantaneous frequencies `frange`, phases `pk`."""

    def __init__(self, ti=None, isSpeech=False, isVoiced=False, a0=None, amplitudes=None, frange=None, pk=None):
        self.ti = [] if ti is None else ti
        self.isSpeech = isSpeech
        self.isVoiced = isVoiced
        self.a0 = [] if a0 is None else a0
        self.ak = [] if amplitudes is None else amplitudes
        self.frange = [] if frange is None else frange
        self.pk = [] if pk is None else pk


class Frame(_Record):
    """One 5 ms voicing-decision frame (functions.py:640)."""

    def __init__(self, ti, isSpeech, isVoiced):
        self.ti = ti
        self.isSpeech = isSpeech
        self.isVoiced = isVoiced
