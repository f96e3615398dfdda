"""Height-field terrain built at set-up time (mirror of reference go1_gym/utils/terrain.py:12-179).

`Terrain` lays sub-terrain tiles into one int16 height field and records the per-tile env origins, exactly like
the reference class (tile grid, borders, the evaluation region behind the training one, `curriculum` / randomised / `selected` modes, the `choice`/`difficulty`
→ generator mapping of `make_terrain` :114-159).  The sub-terrain *generators* themselves live in the closed
`isaacgym.terrain_utils` package, which is not part of the reference tree; the ones below are re-derived from their
documented behaviour (parameters have the same names and units: metres, with `horizontal_scale` /
`vertical_scale` quantisation), not transcribed — exact sample-for-sample agreement with Isaac Gym's versions is
therefore not claimed (DESIGN.md "Terrain").

The physics consumes the height field directly (bilinear surface, csrc/go1_physics.h `terrain_sample`); the
reference's `trimesh` mode additionally turns slopes above `slope_treshold` into vertical walls — not reproduced.
"""
import numpy as np


class SubTerrain:
    """One tile: int16 `height_field_raw` (width x length samples) + the two scales (metres per unit)."""

    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None, rng=np.random):
    """Heights drawn uniformly from {min_height, min_height+step, ..., max_height} [m] on a coarse grid of pitch
    `downsampled_scale` [m], bilinearly interpolated to the tile resolution and ADDED to the tile."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo, hi, st = (int(round(v / terrain.vertical_scale)) for v in (min_height, max_height, step))
    st = max(st, 1)
    levels = np.arange(lo, hi + st, st) if hi > lo else np.array([lo])
    nx = max(int(terrain.width * terrain.horizontal_scale / downsampled_scale), 2)
    ny = max(int(terrain.length * terrain.horizontal_scale / downsampled_scale), 2)
    coarse = rng.choice(levels, (nx, ny)).astype(float)
    gx = np.linspace(0, nx - 1, terrain.width)
    gy = np.linspace(0, ny - 1, terrain.length)
    x0 = np.clip(np.floor(gx).astype(int), 0, nx - 2)
    y0 = np.clip(np.floor(gy).astype(int), 0, ny - 2)
    ax, ay = (gx - x0)[:, None], (gy - y0)[None, :]
    c = coarse
    z = (c[x0][:, y0] * (1 - ax) * (1 - ay) + c[x0 + 1][:, y0] * ax * (1 - ay)
         + c[x0][:, y0 + 1] * (1 - ax) * ay + c[x0 + 1][:, y0 + 1] * ax * ay)
    terrain.height_field_raw += np.rint(z).astype(np.int16)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    """A pyramid whose faces have gradient `slope` (negative: a pit), flat square platform of side
    `platform_size` [m] on top; added to the tile."""
    w, l = terrain.width, terrain.length
    x = np.arange(w)[:, None]
    y = np.arange(l)[None, :]
    cx, cy = w // 2, l // 2
    rise = ((cx - np.abs(cx - x)) / cx) * ((cy - np.abs(cy - y)) / cy)       # 0 at the rim, 1 at the centre
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (w / 2))
    field = max_height * rise
    half = int(platform_size / terrain.horizontal_scale / 2)
    top = field[cx - half, cy - half] if half < min(cx, cy) else field[0, 0]
    field = np.minimum(field, top) if slope >= 0 else np.maximum(field, top)
    terrain.height_field_raw += field.astype(np.int16)
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    """Concentric square steps of tread `step_width` [m] and riser `step_height` [m] (negative: descending towards
    the centre) around a central platform; overwrites the tile."""
    sw = max(int(step_width / terrain.horizontal_scale), 1)
    sh = int(step_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    x0, x1, y0, y1 = 0, terrain.width, 0, terrain.length
    h = 0
    field = terrain.height_field_raw
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0 += sw; x1 -= sw; y0 += sw; y1 -= sw
        h += sh
        field[x0:x1, y0:y1] = h
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0, rng=np.random):
    """`num_rects` axis-aligned boxes with side in [min_size, max_size] [m] and height in
    {-max, -max/2, max/2, max} [m] at random places; flat central platform; overwrites the tile."""
    mh = int(max_height / terrain.vertical_scale)
    lo, hi = int(min_size / terrain.horizontal_scale), int(max_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    i, j = terrain.height_field_raw.shape
    heights = [-mh, -mh // 2, mh // 2, mh]
    sizes = range(max(lo, 1), max(hi, lo + 1), 4)
    for _ in range(num_rects):
        w, l = rng.choice(sizes), rng.choice(sizes)
        a = rng.choice(range(0, max(i - w, 1), 4))
        b = rng.choice(range(0, max(j - l, 1), 4))
        terrain.height_field_raw[a:a + w, b:b + l] = rng.choice(heights)
    x0, y0 = (i - plat) // 2, (j - plat) // 2
    terrain.height_field_raw[x0:x0 + plat, y0:y0 + plat] = 0
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10, rng=np.random):
    """Square stones of side `stone_size` [m] separated by gaps `stone_distance` [m] of depth `depth` [m], stone
    tops jittered within +-`max_height` [m]; flat central platform; overwrites the tile."""
    ss = max(int(stone_size / terrain.horizontal_scale), 1)
    sd = max(int(stone_distance / terrain.horizontal_scale), 1)
    mh = int(max_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    field = terrain.height_field_raw
    field[:, :] = int(depth / terrain.vertical_scale)
    levels = np.arange(-mh - 1, mh, 1) if mh > 0 else np.array([0])
    for a in range(0, terrain.width, ss + sd):
        for b in range(0, terrain.length, ss + sd):
            field[a:a + ss, b:b + ss] = rng.choice(levels)
    x0, y0 = (terrain.width - plat) // 2, (terrain.length - plat) // 2
    field[x0:x0 + plat, y0:y0 + plat] = 0
    return terrain


class Terrain:
    def __init__(self, cfg, num_robots, eval_cfg=None, num_eval_robots=0):
        """reference terrain.py:13-54: the training tile grid, and — with `eval_cfg` — a second grid for the evaluation
        environments appended along x (rows) in the same height field: `x_offset` = the training region's height in samples,
        columns = the wider of the two."""
        self.cfg, self.eval_cfg, self.num_robots = cfg, eval_cfg, num_robots
        self.type = cfg.mesh_type
        if self.type in ("none", "plane"):
            return
        self.train_rows, self.train_cols, self.eval_rows, self.eval_cols = self.load_cfgs()
        self.tot_rows = len(self.train_rows) + len(self.eval_rows)
        self.tot_cols = max(len(self.train_cols), len(self.eval_cols))
        cfg.env_length, cfg.env_width = cfg.terrain_length, cfg.terrain_width
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        self.initialize_terrains()
        self.heightsamples = self.height_field_raw

    def load_cfgs(self):
        """grid sizes of both regions and where the evaluation one starts (reference :37-54)"""
        cfg, eval_cfg = self.cfg, self.eval_cfg
        self._load_cfg(cfg)
        cfg.row_indices = np.arange(0, cfg.tot_rows)
        cfg.col_indices = np.arange(0, cfg.tot_cols)
        cfg.x_offset = 0
        cfg.rows_offset = 0
        if eval_cfg is None:
            return cfg.row_indices, cfg.col_indices, [], []
        self._load_cfg(eval_cfg)
        eval_cfg.row_indices = np.arange(cfg.tot_rows, cfg.tot_rows + eval_cfg.tot_rows)
        eval_cfg.col_indices = np.arange(0, eval_cfg.tot_cols)
        eval_cfg.x_offset = cfg.tot_rows
        eval_cfg.rows_offset = cfg.num_rows
        return cfg.row_indices, cfg.col_indices, eval_cfg.row_indices, eval_cfg.col_indices

    @staticmethod
    def _load_cfg(cfg):
        cfg.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        cfg.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        cfg.width_per_env_pixels = int(cfg.terrain_length / cfg.horizontal_scale)
        cfg.length_per_env_pixels = int(cfg.terrain_width / cfg.horizontal_scale)
        cfg.border = int(cfg.border_size / cfg.horizontal_scale)
        cfg.tot_cols = int(cfg.num_cols * cfg.width_per_env_pixels) + 2 * cfg.border
        cfg.tot_rows = int(cfg.num_rows * cfg.length_per_env_pixels) + 2 * cfg.border

    def initialize_terrains(self):
        self._initialize_terrain(self.cfg)
        if self.eval_cfg is not None:
            self._initialize_terrain(self.eval_cfg)

    def _initialize_terrain(self, cfg):
        """reference :68-74: curriculum beats `selected` beats randomised"""
        if cfg.curriculum:
            self.curriculum(cfg)
        elif cfg.selected:
            self.selected_terrain(cfg)
        else:
            self.randomized_terrain(cfg)

    def randomized_terrain(self, cfg):
        """reference :76-85: a random type and one of three difficulties per tile"""
        for k in range(cfg.num_sub_terrains):
            i, j = k // cfg.num_cols, k % cfg.num_cols
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9])
            self.add_terrain_to_map(cfg, self.make_terrain(cfg, choice, difficulty, cfg.proportions), i, j)

    def curriculum(self, cfg):
        """reference :87-95: difficulty grows with the row, the type follows the column"""
        for j in range(cfg.num_cols):
            for i in range(cfg.num_rows):
                self.add_terrain_to_map(cfg, self.make_terrain(cfg, j / cfg.num_cols + 0.001, i / cfg.num_rows * cfg.difficulty_scale,
                                                               cfg.proportions), i, j)

    def selected_terrain(self, cfg):
        """reference :97-112: one named generator with `terrain_kwargs` on every tile"""
        kind = cfg.terrain_kwargs.pop('type')
        gen = {f.__name__: f for f in (random_uniform_terrain, pyramid_sloped_terrain, pyramid_stairs_terrain,
                                       discrete_obstacles_terrain, stepping_stones_terrain)}[kind.split(".")[-1]]
        for k in range(cfg.num_sub_terrains):
            i, j = k // cfg.num_cols, k % cfg.num_cols
            tile = self._blank(cfg)
            gen(tile, **cfg.terrain_kwargs.terrain_kwargs)
            self.add_terrain_to_map(cfg, tile, i, j)

    @staticmethod
    def _blank(cfg):
        return SubTerrain("terrain", width=cfg.width_per_env_pixels, length=cfg.width_per_env_pixels,
                          vertical_scale=cfg.vertical_scale, horizontal_scale=cfg.horizontal_scale)

    def make_terrain(self, cfg, choice, difficulty, proportions):
        """choice / difficulty -> generator and parameters: the mapping of reference terrain.py:114-159."""
        tile = self._blank(cfg)
        p = list(proportions) + [np.inf] * 10
        slope = difficulty * 0.4
        step_height = 0.05 + 0.18 * difficulty
        obstacle_height = 0.05 + difficulty * (cfg.max_platform_height - 0.05)
        stone_size = 1.5 * (1.05 - difficulty)
        stone_distance = 0.05 if difficulty == 0 else 0.1
        if choice < p[0]:
            pyramid_sloped_terrain(tile, slope=-slope if choice < p[0] / 2 else slope, platform_size=3.)
        elif choice < p[1]:
            pyramid_sloped_terrain(tile, slope=slope, platform_size=3.)
            random_uniform_terrain(tile, min_height=-0.05, max_height=0.05, step=self.cfg.terrain_smoothness, downsampled_scale=0.2)
        elif choice < p[3]:
            pyramid_stairs_terrain(tile, step_width=0.31, step_height=-step_height if choice < p[2] else step_height, platform_size=3.)
        elif choice < p[4]:
            discrete_obstacles_terrain(tile, obstacle_height, 1., 2., 20, platform_size=3.)
        elif choice < p[5]:
            stepping_stones_terrain(tile, stone_size=stone_size, stone_distance=stone_distance, max_height=0., platform_size=4.)
        elif choice < p[7]:
            pass
        elif choice < p[8]:
            if cfg.terrain_noise_magnitude != 0:
                random_uniform_terrain(tile, min_height=-cfg.terrain_noise_magnitude, max_height=cfg.terrain_noise_magnitude,
                                       step=0.005, downsampled_scale=0.2)
        elif choice < p[9]:
            random_uniform_terrain(tile, min_height=-0.05, max_height=0.05, step=self.cfg.terrain_smoothness, downsampled_scale=0.2)
            tile.height_field_raw[0:tile.length // 2, :] = 0
        return tile

    def add_terrain_to_map(self, cfg, terrain, row, col):
        sx = cfg.border + row * cfg.length_per_env_pixels + cfg.x_offset
        sy = cfg.border + col * cfg.width_per_env_pixels
        self.height_field_raw[sx:sx + cfg.length_per_env_pixels, sy:sy + cfg.width_per_env_pixels] = terrain.height_field_raw
        # spawn height of the tile.  Reference terrain.py:177: the maximum over the WHOLE tile — the 2 x 2 m centre patch of
        # legged_gym is computed at :173-176 and left unused, so on a pit-shaped tile (inverted slope, descending stairs) a reset
        # drops the robot from the rim's height onto the centre platform.  Mirrored as the default; the non-reference switch
        # `terrain.origin_height_source = "centre_patch"` takes the patch instead (profiles/r04_rough_train_sanity.txt uses it to
        # separate this spawn rule from the simulator in BASELINE configs[2]'s learning curve).
        if getattr(cfg, "origin_height_source", "tile_max") == "centre_patch":
            cx, cy = sx + cfg.length_per_env_pixels // 2, sy + cfg.width_per_env_pixels // 2
            hx = max(1, int(1.0 / terrain.horizontal_scale))
            z = np.max(self.height_field_raw[cx - hx:cx + hx, cy - hx:cy + hx]) * terrain.vertical_scale
        else:
            z = np.max(self.height_field_raw[sx:sx + cfg.length_per_env_pixels, sy:sy + cfg.width_per_env_pixels]) * terrain.vertical_scale
        cfg.env_origins[row, col] = [(row + 0.5) * cfg.terrain_length + cfg.x_offset * terrain.horizontal_scale,
                                     (col + 0.5) * cfg.terrain_width, z]
