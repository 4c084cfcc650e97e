"""Host-side mirror of brevitas.core: the modules a resolved quantizer graph is made of."""
