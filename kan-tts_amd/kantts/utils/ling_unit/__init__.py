"""Linguistic symbol tables of the acoustic model (reference kantts/utils/ling_unit/): the integer ids SAM-BERT's
embeddings are indexed with.  Native here so that SAM-BERT training / inference on real data does not need the reference
package; the language RESOURCE files (PhoneSet.xml, tonelist.txt of a language) are data, not code, and are read from a
directory the user points to (see ling_unit.language_directory).  The text front-end (ttsfrd: text -> symbols) is NOT
part of this package."""
from kantts.utils.ling_unit.ling_unit import (EMOTION_TYPES, SYLLABLE_FLAGS, WORD_SEGMENTS, KanTtsLinguisticUnit,  # noqa: F401
                                              language_directory, load_language_symbols)
