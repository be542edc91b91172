"""Host-side mirror of the reference's `model` package (same class names, constructor
signatures and attribute / state_dict names), executing on libodamd.so."""
