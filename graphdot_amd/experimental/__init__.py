"""Import paths of the reference's experimental namespace."""
