// hand-written parser fixture: every object kind and modifier the subset knows
// second comment line
camera {
   location <1.5, 2.25, -9.75>
   up <0, 1, 0>
   right <1.7778, 0, 0>
   look_at <0.25, 0.5, 0>
}

light_source {<-30.5, 40, 12> color rgb <1.25, 1.0, 0.75>}
light_source { < 5 , 6 , 7 > color rgb < 0.1 , 0.2 , 0.3 > }

area_light {<-1, 5, -1> <1, 5, -1> <1, 5, 1> sample 3 color rgb <0.9, 0.9, 0.9>}

sphere { <0.5, -1, 2>, 1.75
   pigment { color rgb <0.3, 0.4, 0.5>}
   finish {ambient 0.15 diffuse 0.55 specular 0.7 roughness 0.02 reflection 0.35}
   scale <1.5, 0.5, 2>
   rotate <30, -45, 60>
   translate <-2, 0.125, 3.5>
}

sphere {<3, 3, 3>, 0.5
  translate <0, 0, -0>
  pigment { color rgbf <1, 0.9, 0.7, 0.65> }
  finish { ambient 0.125 diffuse 0.5 refraction 1.0 ior 1.33}
}

box { <-1, -2, -3>, <4, 0.5, 1>
   pigment { color rgb <0.9, 0.1, 0.1, 0.7>}
   finish { ambient 0.3 }
   rotate <0, 33.3, 0>
}

cone { <0, 0, 0>, 1.0, <0, 2, 0>, 0.25
   pigment { color rgb <0, 1, 0>}
}

plane {<0, 1, 0>, -4.5
   pigment {color rgb <0.2, 0.2, 0.8>}
   finish {ambient 0.4 diffuse 0.8}
   rotate <0, 0, 12.5>
}

triangle {
  <-0.5, 0, 0.25>,
  <0.5, 0.125, 0>,
  <0, 1.5, -0.75>
  pigment { color rgb <0.4, 0.6, 0.4>}
  finish {ambient 0.2 diffuse 0.6 specular 0.5 roughness 10.0}
  scale <2, 2, 2>
}

smooth_triangle {
  <0, 0, 0>, <0, 0, 1>,
  <1, 0, 0.1>, <0.1, 0, 0.99>,
  <0, 1, 0.2>, <0, 0.1, 0.99>
  pigment { color rgbf <1.0, 1.0, 1.0, 0.8> }
  finish { ambient 0.1 diffuse 0.2 reflection 0.3 refraction 1.0 ior 1.5 }
  translate <0.5, 0.5, 0.5>
  rotate <10, 20, 30>
}
