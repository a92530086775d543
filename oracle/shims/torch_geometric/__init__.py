"""Test-infrastructure stub for the absent `torch_geometric`."""
