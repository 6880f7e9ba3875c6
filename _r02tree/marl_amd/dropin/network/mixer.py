from marl_amd.network.mixer import VDNMixer, QMixMixer, DMAQer, DMAQ_SI_Weight, QtranQBase, QtranQAlt, QtranV  # noqa: F401
