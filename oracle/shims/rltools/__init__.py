"""Stand-in for the un-vendored `rltools` submodule (empty in /root/reference)."""
