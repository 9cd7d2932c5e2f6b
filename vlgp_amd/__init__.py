"""vlgp_amd -- MI355X-native variational-EM engine for vLGP (drop-in for the
``vlgp.fit`` hot path of catniplab/vlgp)."""
