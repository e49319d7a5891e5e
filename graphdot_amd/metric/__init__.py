"""Graph distance metrics built on the marginalized graph kernel."""
