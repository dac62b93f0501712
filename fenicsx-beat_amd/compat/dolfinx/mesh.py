from beat.grid import (  # noqa: F401
    CellType, Mesh, MeshTags, create_box, create_interval, create_rectangle, create_unit_cube, create_unit_interval,
    create_unit_square, locate_entities, locate_entities_boundary, meshtags,
)
