"""Test-infrastructure stub for the absent `pykeops` (only Biasutti/No3D use it)."""
