"""Stand-in for the `smac` package (StarCraft II is not vendored by the reference): `smac.env.StarCraft2Env` is the
synthetic SMAC-shaped device env.  On sys.path only when the real `smac` is not importable (marl_amd.dropin launcher)."""
